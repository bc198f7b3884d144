cd /root/repo
python -m pytest tests -x -q -m gpu -k "wgrad" 2>&1 | tail -3
for v in 0 1; do for sh in "9 1024 8 16 u" "9 1024 8 16" "9 1024 8 8" "3 1024 8 8" "3 1024 16 8" "9 512 16 32"; do echo -n "fix $v: $sh: "; PG_WGRAD_THIN_FIX=$v python tools/exp/one_wgrad.py $sh 2>&1 | grep us; done; done
B="python bench.py --no-cpu --no-per-depth --no-configs --no-kernel-timing --steps 30 --warmup 5"
for v in 0 1 0 1; do PG_WGRAD_THIN_FIX=$v $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fix $v', d['value'], d['ms_per_step'])"; done
