cd /root/repo
python -m pytest tests/test_winograd.py -x -q -m gpu -k "pixelnorm or engine" 2>&1 | tail -5
B="python bench.py --no-cpu --no-per-depth --no-configs --no-kernel-timing --steps 20 --warmup 5"
for i in 1 2; do $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fused-pn', d['value'], d['ms_per_step'])"; done
python -m pytest tests/test_e2e_gpu.py tests/test_engine_gpu.py -x -q -m gpu 2>&1 | tail -3
