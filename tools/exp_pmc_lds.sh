cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in buf5 swz; do
  PGGAN_HIP_LIB=$R/ab/lib_$v.so timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -d $R/gpurun_out/pmclds_$v -o p --output-format csv -- python $R/tools/sweep_wino.py > $R/gpurun_out/pmclds_$v.log 2>&1
  python - <<PY
import csv,glob,collections
f=glob.glob('$R/gpurun_out/pmclds_$v/**/p_counter_collection.csv',recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    import re
    m=re.search(r'(conv_\w+<[^>]*>|conv_\w+)', r['Kernel_Name']); k=m.group(1) if m else r['Kernel_Name'][:40]
    acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
for k,d in acc.items():
    if 'wino_kernel' in k or 'igemm' in k:
        print('$v',k,'conflict/active = %.3f'%(d['SQ_LDS_BANK_CONFLICT']/max(1,d['SQ_LDS_IDX_ACTIVE'])), 'active %.3g'%d['SQ_LDS_IDX_ACTIVE'])
PY
  rm -rf $R/gpurun_out/pmclds_$v
done
