for r in 1 2 4 8 1 2 4 8; do echo "== PG_WSTRIP_ROUNDS=$r"; PG_WSTRIP_ROUNDS=$r python bench.py --no-per-depth --no-configs --no-cpu --steps 40 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print(round(d['value'],1), round(d['ms_per_step'],3), 'strip ms/step', round(sum(v['ms_per_step'] for n,v in k.items() if 'wino_strip' in n),3), {n[22:]:round(v['avg_launch_us']) for n,v in k.items() if 'wino_strip' in n})"; done
