"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, counters summed over dispatches,
plus derived ratios (MFMA busy fraction from dispatch timestamps at 2.4 GHz x 1024 SIMDs)."""
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(dict)
for path in sys.argv[1:]:
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row['Kernel_Name']
            k = k.replace('(anonymous namespace)::', '').replace('void ', '')
            k = k.split('(')[0][:44]
            agg[k][row['Counter_Name']] += float(row['Counter_Value'])
            dur[k][(path, row['Dispatch_Id'])] = int(row['End_Timestamp']) - int(row['Start_Timestamp'])
names = sorted({c for k in agg for c in agg[k]})
tot = {k: sum(d.values()) for k, d in dur.items()}
npass = len(sys.argv) - 1
print('%-46s %6s %9s %7s | %s' % ('kernel', 'calls', 'total_ms', 'mfma%', ' '.join('%12s' % n.replace('SQ_', '')[-12:] for n in names)))
for k in sorted(agg, key=lambda k: -tot[k]):
    d = agg[k]
    t_ns = tot[k] / npass
    mf = 100.0 * d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (t_ns * 2.4 * 1024) if t_ns else 0
    print('%-46s %6d %9.3f %7.1f | %s' % (k, len(dur[k]) // npass, t_ns / 1e6, mf, ' '.join('%12.4g' % d.get(n, 0) for n in names)))
