cd /root/repo
B="python bench.py --no-cpu --no-per-depth --no-configs --no-kernel-timing --steps 30 --warmup 5"
for v in "" "3=9" "" "3=9"; do PGGAN_TUNE=$v $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tune [$v]', d['value'], d['ms_per_step'])"; done
