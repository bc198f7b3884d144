cd /root/repo
python -m pytest tests -x -q -m gpu -k "conv and not wino" 2>&1 | tail -2
B="python bench.py --no-cpu --no-per-depth --no-configs --no-kernel-timing --steps 30 --warmup 5"
for lib in "" ab/libpggan_kcp20.so ab/libpggan_kcp28.so "" ab/libpggan_kcp20.so ab/libpggan_kcp28.so; do PGGAN_HIP_LIB=$lib $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib [$lib]', d['value'], d['ms_per_step'])"; done
