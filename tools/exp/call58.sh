cd /root/repo
python tools/exp/smallm_bench.py 2>&1 | grep -v amdgpu
B="python bench.py --no-cpu --no-per-depth --no-configs --no-kernel-timing --steps 30 --warmup 5"
for v in 0 1 0 1; do PGGAN_SMALLM=$v $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('smallm $v', d['value'], d['ms_per_step'])"; done
