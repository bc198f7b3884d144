"""Condense the rocprofv3 outputs of tools/profile_round.sh into small committed summaries:
   <out>/<tag>_kernel_stats.csv   (rocprofv3 --kernel-trace --stats, verbatim per-kernel table)
   <out>/<tag>_pmc_summary.csv    (per kernel: calls, total ms, HBM bytes/launch, MFMA busy %, LDS conflicts)
   <out>/<tag>_roofline.json      (what bench.py reads for roofline.traffic)
HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are in KiB,
collected in SEPARATE passes, and on gfx950 FETCH_SIZE under-reports wide coalesced reads by exactly 2x."""
import collections
import csv
import json
import os
import shutil
import sys

out, tag = sys.argv[1], sys.argv[2]


def short(k):
    k = k.replace('(anonymous namespace)::', '').replace('void ', '')
    return k.split('(')[0]


def load_pmc(path, window=False):
    """window: keep only the dispatches between the two marker launches (minmax_kernel) bench.py --d-step-only puts around its timed
    steps -- the kernels of the window, not those of the set-up (network construction, first derivation of the weights, warm-up)."""
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(dict)
    if not os.path.exists(path):
        return agg, disp
    lo = hi = None
    if window:
        with open(path) as f:
            marks = sorted({int(row['Dispatch_Id']) for row in csv.DictReader(f) if 'minmax_kernel' in row['Kernel_Name']})
        if len(marks) >= 2:
            lo, hi = marks[0], marks[-1]
        else:
            print('load_pmc: no window markers in %s (whole process summed)' % path)
    with open(path) as f:
        for row in csv.DictReader(f):
            if lo is not None and not (lo < int(row['Dispatch_Id']) < hi):
                continue
            k = short(row['Kernel_Name'])
            agg[k][row['Counter_Name']] += float(row['Counter_Value'])
            disp[k][row['Dispatch_Id']] = int(row['End_Timestamp']) - int(row['Start_Timestamp'])
    return agg, disp


ks = os.path.join(out, 'kt', 'kt_kernel_stats.csv')
if os.path.exists(ks):
    shutil.copy(ks, os.path.join(out, tag + '_kernel_stats.csv'))
fetch, fd = load_pmc(os.path.join(out, 'pmc_fetch', 'p_counter_collection.csv'))
write, wd = load_pmc(os.path.join(out, 'pmc_write', 'p_counter_collection.csv'))
sq, sd = load_pmc(os.path.join(out, 'pmc_sq', 'p_counter_collection.csv'))
lds, ld = load_pmc(os.path.join(out, 'pmc_lds', 'p_counter_collection.csv'))
dsq, dsd = load_pmc(os.path.join(out, 'pmc_sq_dstep', 'p_counter_collection.csv'), window=True)     # bench.py --d-step-only: the D step + gradient penalty window
kernels = sorted(set(fetch) | set(write) | set(sq), key=lambda k: -sum(sd.get(k, fd.get(k, {})).values()))
rows = []
roof = {}
for k in kernels:
    calls = len(sd.get(k) or fd.get(k) or wd.get(k) or {})
    tot_ns = sum((sd.get(k) or fd.get(k) or {}).values())
    fcalls, wcalls = max(1, len(fd.get(k, {}))), max(1, len(wd.get(k, {})))
    fetch_b = 2.0 * 1024.0 * fetch.get(k, {}).get('FETCH_SIZE', 0.0) / fcalls        # gfx950 x2 correction
    write_b = 1024.0 * write.get(k, {}).get('WRITE_SIZE', 0.0) / wcalls
    mf = sq.get(k, {}).get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
    t_sq = sum(sd.get(k, {}).values())
    mfma_pct = 100.0 * mf / (t_sq * 2.4 * 1024) if t_sq else 0.0
    conf = lds.get(k, {}).get('SQ_LDS_BANK_CONFLICT', 0.0)
    act = lds.get(k, {}).get('SQ_LDS_IDX_ACTIVE', 0.0)
    rows.append([k, calls, '%.3f' % (tot_ns / 1e6), '%.1f' % (tot_ns / 1e3 / max(1, calls)), '%.0f' % fetch_b, '%.0f' % write_b,
                 '%.1f' % mfma_pct, '%.3f' % (conf / act if act else 0.0),
                 '%.3g' % sq.get(k, {}).get('SQ_INSTS_MFMA', 0), '%.3g' % sq.get(k, {}).get('SQ_INSTS_VALU', 0),
                 '%.3f' % (sq.get(k, {}).get('SQ_WAIT_INST_ANY', 0) / max(1.0, sq.get(k, {}).get('SQ_WAVE_CYCLES', 1))),
                 '%.3f' % (sq.get(k, {}).get('SQ_WAIT_ANY', 0) / max(1.0, sq.get(k, {}).get('SQ_WAVE_CYCLES', 1))),
                 # effective shader clock while the kernel ran: GRBM_GUI_ACTIVE cycles / kernel wall time (MI355X_MICROARCH.md, DVFS).  rocprofv3
                 # sums the counter over the 8 XCDs; it also counts the dispatch time around a kernel, so launches shorter than ~100 us read
                 # high (an upper bound there; the 200-330 us launches read 2.25 GHz)
                 '%.2f' % (lds.get(k, {}).get('GRBM_GUI_ACTIVE', 0.0) / 8.0 / max(1.0, sum(ld.get(k, {}).values()))),
                 # round 4: what the waves wait for.  VALU instructions per MFMA (fp32 MFMA time and VALU time of the waves of a SIMD ADD on
                 # gfx950, tools/exp/mfma_valu_share.hip: MFMA-busy <= 32 / (32 + 4 x this)); LDS pipe active, waves waiting for LDS and
                 # waves executing VALU / LDS instructions as fractions of the wave cycles of the same pass
                 '%.2f' % (sq.get(k, {}).get('SQ_INSTS_VALU', 0) / sq.get(k, {}).get('SQ_INSTS_MFMA', 1) - 1.0 if sq.get(k, {}).get('SQ_INSTS_MFMA', 0) else 0.0),
                 '%.3f' % (lds.get(k, {}).get('SQ_LDS_IDX_ACTIVE', 0) / max(1.0, lds.get(k, {}).get('SQ_WAVE_CYCLES', 0) or 1e30)),
                 '%.3f' % (lds.get(k, {}).get('SQ_WAIT_INST_LDS', 0) / max(1.0, lds.get(k, {}).get('SQ_WAVE_CYCLES', 0) or 1e30)),
                 '%.3f' % (lds.get(k, {}).get('SQ_ACTIVE_INST_VALU', 0) / max(1.0, lds.get(k, {}).get('SQ_WAVE_CYCLES', 0) or 1e30)),
                 '%.3f' % (lds.get(k, {}).get('SQ_ACTIVE_INST_LDS', 0) / max(1.0, lds.get(k, {}).get('SQ_WAVE_CYCLES', 0) or 1e30))])
    nm, nv = sq.get(k, {}).get('SQ_INSTS_MFMA', 0), sq.get(k, {}).get('SQ_INSTS_VALU', 0)
    roof[k] = {'hbm_bytes_per_launch': fetch_b + write_b, 'fetch_bytes_per_launch': fetch_b,
               'write_bytes_per_launch': write_b, 'mfma_busy_pct': mfma_pct, 'avg_launch_us': tot_ns / 1e3 / max(1, calls),
               'valu_per_mfma': (nv / nm - 1.0) if nm else None}       # (SQ_INSTS_VALU counts the MFMAs too)
with open(os.path.join(out, tag + '_pmc_summary.csv'), 'w') as f:
    w = csv.writer(f)
    w.writerow(['kernel', 'calls', 'total_ms', 'avg_us', 'hbm_fetch_bytes_per_launch(x2 corrected)', 'hbm_write_bytes_per_launch',
                'mfma_busy_pct', 'lds_conflict_frac', 'insts_mfma', 'insts_valu', 'wait_inst_frac', 'wait_any_frac', 'eff_clock_ghz',
                'valu_per_mfma', 'lds_idx_active_frac', 'wait_inst_lds_frac', 'active_inst_valu_frac', 'active_inst_lds_frac'])
    w.writerows(rows)
fam = collections.defaultdict(lambda: dict(bytes=0.0, calls=0, ns=0.0))
for k, v in roof.items():
    base = k.split('<')[0]
    n = len(sd.get(k) or fd.get(k) or {})
    fam[base]['bytes'] += v['hbm_bytes_per_launch'] * n
    fam[base]['calls'] += n
# ---- the north-star window by counter: D step + gradient penalty + Adam(D) only (bench.py --d-step-only under --pmc)
DSTEP_PASSES = int(os.environ.get('PG_DSTEP_PASSES', '3'))      # profile_round.sh: --warmup 3 --steps 3 for the --d-step-only pass
dwin = None
if dsq:
    tot = sum(sum(v.values()) for v in dsd.values())
    conv = [k for k in dsd if k.startswith(('conv_', 'wgrad_strip'))]
    def busy(keys):
        t = sum(sum(dsd[k].values()) for k in keys)
        return 100.0 * sum(dsq[k].get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) for k in keys) / (t * 2.4 * 1024) if t else 0.0
    dwin = {'mfma_busy_pct_all_kernels': busy(list(dsd)), 'mfma_busy_pct_conv_kernels': busy(conv),
            'conv_kernel_time_share': sum(sum(dsd[k].values()) for k in conv) / tot if tot else 0.0,
            'kernel_time_ms_per_pass': tot / 1e6 / DSTEP_PASSES, 'passes': DSTEP_PASSES,
            'is': 'SQ_VALU_MFMA_BUSY_CYCLES / (kernel time x 2.4 GHz x 1024 SIMDs), summed over the kernels launched between the two window markers of bench.py --d-step-only (its timed passes, serialised by the counter collection)'}
    print('D step + GP window:', dwin)
json.dump({'tag': tag, 'd_step_gp_window': dwin, 'per_kernel': roof,
           'per_family': {b: {'hbm_bytes_per_launch': d['bytes'] / max(1, d['calls']), 'launches': d['calls']} for b, d in fam.items()}},
          open(os.path.join(out, tag + '_roofline.json'), 'w'), indent=1)
for r in rows[:14]:
    print(r)

# ---- bench.py <-> rocprofv3 attribution check: every conv symbol the bench line times must be launched exactly as often per
# step as the kernel trace of the same workload says (a wrapper missing from bench.KernelTimer shows up here)
bench_json = os.path.join(out, 'bench_detail.json')          # bench.py's detail file of the un-traced run (the stdout line carries no tables)
steps_traced = int(os.environ.get('PG_TRACED_STEPS', '23'))            # profile_round.sh: --prime 10 --warmup 3 --steps 10
if os.path.exists(bench_json) and os.path.exists(ks):
    bk = json.load(open(bench_json)).get('kernels', {})
    traced = {}
    with open(ks) as f:
        for row in csv.DictReader(f):
            traced[short(row['Name'])] = int(row['Calls']) / float(steps_traced)
    bad = []
    conv_syms = [k for k in traced if k.startswith(('conv_', 'wgrad_strip')) and 'epilogue' not in k]
    for k in sorted(set(bk) | set(conv_syms)):
        kb = k if k in traced else next((t for t in traced if t.startswith(k.rstrip('>') + ',') or t == k), None)    # (a trailing template flag the bench name omits)
        b, r = bk.get(k, {}).get('launches_per_step', 0.0), traced.get(kb, 0.0) if kb else 0.0
        if k not in bk and any(k.startswith(x.rstrip('>') + ',') for x in bk):
            continue
        # (tolerance: the first traced step has no early real-third pass yet -- Trainer starts it for the NEXT iteration -- so a symbol that
        #  pass launches k times reads k / steps_traced low)
        print('%-58s bench %6.2f  rocprofv3 %6.2f launches per step%s' % (k, b, r, '' if abs(b - r) < 0.25 else '   <-- MISMATCH'))
        if abs(b - r) >= 0.25:
            bad.append(k)
    if bad:
        print('ATTRIBUTION MISMATCH for: %s' % ', '.join(bad))
        sys.exit(1)
    print('attribution check: %d conv symbols agree between bench.py and rocprofv3' % len(set(bk) | set(conv_syms)))
