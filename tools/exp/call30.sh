cd /root/repo
python -m pytest tests/test_winograd.py -x -q -m gpu -k "kernel" 2>&1 | tail -3
B="python bench.py --no-cpu --no-per-depth --no-configs --no-kernel-timing --steps 20 --warmup 5"
for xs in 4 3 4 3; do PG_WINO_XS=$xs $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('xs $xs', d['value'], d['ms_per_step'])"; done
for xs in 4 3; do echo "== XS $xs"; PG_WINO_XS=$xs python tools/sweep_wino.py 2>&1 | sed 's/gen1 [^|]*| //; s/direct [^|]*| //'; done
