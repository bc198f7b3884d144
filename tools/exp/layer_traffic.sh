#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of single Winograd conv layers: bash tools/exp/layer_traffic.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for shape in "9 256 32 32" "9 64 128 256" "9 512 16 16" "9 1024 8 16"; do
  tag=$(echo $shape | tr ' ' '_')
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/one/${tag}_$c -o p --output-format csv -- python $R/tools/exp/one_layer.py $shape > /dev/null 2>&1
  done
  python - <<PY
import csv, glob, collections
tot = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob('$R/gpurun_out/one/${tag}_%s/**/p_counter_collection.csv' % c, recursive=True)[0]
    v, n = 0.0, 0
    for r in csv.DictReader(open(f)):
        if 'wino2' in r['Kernel_Name'] and r['Counter_Name'] == c:
            v += float(r['Counter_Value']); n += 1
    tot[c] = v / max(n, 1)
N, H, ci, co = [int(x) for x in '$shape'.split()]
# MI355X_MICROARCH.md, HBM section: FETCH_SIZE / WRITE_SIZE are in KiB, separate passes; on gfx950 FETCH_SIZE under-reports wide reads by 2x (as tools/summarize_profile.py)
fetch, write = tot['FETCH_SIZE'] * 1024 * 2, tot['WRITE_SIZE'] * 1024
alg = 4.0 * N * H * H * (ci + co)
print('conv n%d @%d %d->%d: HBM fetch %.1f MB + write %.1f MB = %.1f MB per launch; algorithmic (x once + y once) %.1f MB' % (N, H, ci, co, fetch / 1e6, write / 1e6, (fetch + write) / 1e6, alg / 1e6))
PY
  rm -rf $R/gpurun_out/one/${tag}_*
done
