cd /tmp && export TMPDIR=/tmp
R=/root/repo
for shape in "9 1024 8 16 u" "9 1024 8 16" "9 1024 8 8"; do
  tag=$(echo $shape | tr ' ' '_')
  python $R/tools/exp/one_wgrad.py $shape 2>&1 | grep us
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $R/gpurun_out/one/$tag -o p --output-format csv -- python $R/tools/exp/one_wgrad.py $shape > /dev/null 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob('$R/gpurun_out/one/$tag/**/p_counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if 'wgrad' not in k: continue
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
for k, d in acc.items():
    print('$shape', k[:70])
    w = d['SQ_WAVES'] / n[(k, 'SQ_WAVES')]
    for c, v in sorted(d.items()):
        v /= n[(k, c)]
        print('   %-18s %12.0f  per wave %8.1f' % (c, v, v / w))
PY
  rm -rf $R/gpurun_out/one/$tag
done
