cd /root/repo
python -m pytest tests -x -q -m gpu 2>&1 | tail -5
PGGAN_FORCE_DP=1 python bench.py --no-cpu --no-per-depth --no-configs --no-kernel-timing --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dp1', d['value'], d['ms_per_step'], d.get('rccl_ranks'), d.get('allreduce_bytes_per_step'), d.get('allreduce_collectives_per_step'), d.get('exposed_exchange_ms'))"
