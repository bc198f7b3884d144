cd /root/repo
python -m pytest tests/test_winograd.py -x -q -m gpu -k "wino_kernel or pixelnorm" 2>&1 | tail -3
for v in 1 0; do echo "== NG env $v"; PG_WINO_NG=$v python tools/sweep_wino.py 2>&1 | grep -v amdgpu | sed 's/gen1 [^|]*| //; s/direct [^|]*| //; s/gen2\/32 [^|]*| //; s/gen2\/16 [^|]*| //' | head -27; done
B="python bench.py --no-cpu --no-per-depth --no-configs --no-kernel-timing --steps 30 --warmup 5"
for v in 1 0 1 0; do PG_WINO_NG=$v $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ng env $v', d['value'], d['ms_per_step'])"; done
