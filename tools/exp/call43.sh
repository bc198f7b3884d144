cd /root/repo
python -m pytest tests/test_winograd.py -x -q -m gpu -k "wgrad" 2>&1 | tail -3
for v in 0 1; do echo "== PAIR $v"; PG_WW_PAIR=$v python tools/sweep_wino_wgrad.py 2>&1 | grep -v amdgpu | cut -c1-120 | head -12; done
B="python bench.py --no-cpu --no-per-depth --no-configs --no-kernel-timing --steps 20 --warmup 5"
for v in 0 1 0 1; do PG_WW_PAIR=$v $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pair $v', d['value'], d['ms_per_step'])"; done
