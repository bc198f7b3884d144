cd /root/repo
python -m pytest tests/test_winograd.py tests/test_e2e_gpu.py tests/test_checkpoint.py -x -q -m gpu 2>&1 | tail -3
B="python bench.py --no-cpu --no-per-depth --no-configs --no-kernel-timing --steps 30 --warmup 5"
for i in 1 2 3; do $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lazy-wino', d['value'], d['ms_per_step'])"; done
