#!/bin/bash
# usage: ab.sh "<env A>" "<env B>" [pairs] [extra bench.py arguments, e.g. --depth 6]   -- alternating headline runs on the same box
A="$1"; B="$2"; P=${3:-2}; shift; shift; shift
for i in $(seq 1 $P); do for v in "$A" "$B"; do echo -n "[$v] "; env $v timeout 300 python bench.py --no-cpu --no-per-depth --no-configs --no-kernel-timing --steps 40 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['d_step_gp']['ms'])"; done; done
