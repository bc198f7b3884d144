Q="--no-cpu --no-per-depth --no-configs --no-kernel-timing --steps 40 --warmup 10"
for i in 1 2 3; do
echo "== new"; python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
echo "== head"; PGGAN_HIP_LIB=ab/libpggan_head.so python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
done
