cd /root/repo
python -m pytest tests/test_winograd.py -x -q -m gpu -k "wgrad or engine" 2>&1 | tail -4
B="python bench.py --no-cpu --no-per-depth --no-configs --no-kernel-timing --steps 20 --warmup 5"
for d in 0 1 0 1; do PGGAN_DEFER_TANGENT_WGRAD=$d $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('defer $d', d['value'], d['ms_per_step'])"; done
