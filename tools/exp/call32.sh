cd /root/repo
python -m pytest tests/test_winograd.py -x -q -m gpu -k "kernel" 2>&1 | tail -3
B="python bench.py --no-cpu --no-per-depth --no-configs --no-kernel-timing --steps 20 --warmup 5"
for i in 1 2; do $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('peel+nodiv', d['value'], d['ms_per_step'])"; done
python tools/sweep_wino.py 2>&1 | sed 's/gen1 [^|]*| //; s/direct [^|]*| //; s/gen2\/32 [^|]*| //'
bash tools/exp/call31.sh 2>&1 | grep -E "wino2|VALU|SALU|WAVE_CYCLES"
