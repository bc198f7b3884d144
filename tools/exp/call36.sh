cd /root/repo
B="python bench.py --no-cpu --no-per-depth --no-configs --no-kernel-timing --steps 20 --warmup 5"
for t in 512 768 384 640 512 768; do PG_WW_TARGET=$t $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ww target $t', d['value'], d['ms_per_step'])"; done
