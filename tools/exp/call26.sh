cd /root/repo
B="python bench.py --no-cpu --no-per-depth --no-configs --no-kernel-timing --steps 20 --warmup 5"
for c in 32 16 8 32 16 8; do PGGAN_WINO_MIN_C=$c $B 2>/dev/null | tail -1 > gpurun_out/r2_b_minc$c.log; python -c "import json; d=json.loads(open('gpurun_out/r2_b_minc$c.log').read()); print('minc $c', d['value'], d['ms_per_step'])"; done
python -m pytest tests/test_winograd.py tests/test_e2e_gpu.py tests/test_fp64_adjudicator.py -x -q -m gpu 2>&1 | tail -5
