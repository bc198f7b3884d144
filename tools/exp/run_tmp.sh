Q="--no-cpu --no-per-depth --no-configs --no-kernel-timing --steps 40 --warmup 10"
for i in 1 2; do
for t in 512 384 256 768; do
echo "== PG_WW_TARGET=$t"; PG_WW_TARGET=$t python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
done
done
