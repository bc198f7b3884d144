"""Per-layer table of the Winograd conv launches of the 1024x1024 train step ("next round" 2.ii): for every distinct
launch of one real step -- (minibatch, map size, Cin, Cout, epilogue form) as the engine issues it -- the kernel symbol, K chunks,
workgroups, duration, MFMA-busy %, VALU instructions per MFMA and algorithmic / executed TFLOP/s, measured alone on the device on
cold inputs under rocprofv3 --pmc.

  python tools/layer_table.py run  <list.json>      (under rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA)
  python tools/layer_table.py table <list.json> <p_counter_collection.csv> <out.csv>

`run` records the conv2d_wino* calls of ONE train step of the benchmark trainer (shapes and flags only), de-duplicates them and
replays each distinct launch REPS times on rotating synthetic inputs, in a fixed order; `table` joins the per-dispatch counters
(dispatch order = replay order) with that list.  tools/profile_round.sh runs both and copies the CSV next to the PMC summary."""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REPS, ROT = 4, 3


def record_step():
    import torch
    import bench
    import pggan_amd as pg
    ops = pg.ops
    pg.wgan_gp_loss.enable_graphs(False)                        # eager launches: a replayed launch plan does not pass through ops.*
    tr = bench.make_trainer(pg, 1024, 8, 1.0, 3, 1337, None)
    for _ in range(3):
        tr.train()
    calls = collections.OrderedDict()

    def key_of(fn, a, k):
        if fn == 'conv2d_wino':
            x, u, bias, n, h = a[0], a[1], a[2], a[3], a[4]
            mask, upmask = k.get('mask'), k.get('upmask')
            kind = []
            if k.get('ups'): kind.append('ups')
            if bias is not None: kind.append('bias')
            if mask is not None: kind.append('maskb' if mask.dtype == torch.uint8 else 'mask32')
            if k.get('signs_out'): kind.append('signs')
            if k.get('pool'): kind.append('pool' + ('only' if k.get('pool_only') else '') + ('+bytes' if k.get('y_bytes') else '') + ('+blend' if k.get('other') is not None else ''))
            if k.get('unpool'): kind.append('unpool' + ('b' if upmask is not None and upmask.dtype == torch.uint8 else ('32' if upmask is not None else '')))
            return (fn, int(n), int(h), int(u.shape[2]), int(u.shape[1]), '+'.join(kind) or 'plain')
        if fn == 'conv2d_wino_pixelnorm':
            u, n, h = a[1], a[3], a[4]
            return (fn, int(n), int(h), int(u.shape[2]), int(u.shape[1]), 'ups+pn' if k.get('ups') else 'pn')
        u, n, h = a[1], a[4], a[5]
        return (fn, int(n), int(h), int(u.shape[2]), int(u.shape[1]), 'pnbwd+pool' if k.get('pool') else 'pnbwd')
    saved = {}
    for fn in ('conv2d_wino', 'conv2d_wino_pixelnorm', 'conv2d_wino_pnbwd'):
        orig = saved[fn] = getattr(ops, fn)

        def wrapped(*a, _fn=fn, _orig=orig, **k):
            kk = key_of(_fn, a, k)
            calls[kk] = calls.get(kk, 0) + 1
            return _orig(*a, **k)
        setattr(ops, fn, wrapped)
    tr.train()
    torch.cuda.synchronize()
    for fn, orig in saved.items():
        setattr(ops, fn, orig)
    del tr
    torch.cuda.empty_cache()
    return [dict(fn=k[0], n=k[1], H=k[2], cin=k[3], cout=k[4], kind=k[5], launches_per_step=v) for k, v in calls.items()]


def replay(layers):
    import torch
    import pggan_amd as pg
    ops, lib = pg.ops, pg._lib.load()
    g = torch.Generator(device='cuda').manual_seed(1)
    for L in layers:
        n, H, ci, co, kind = L['n'], L['H'], L['cin'], L['cout'], L['kind']
        ups = 'ups' in kind.split('+')
        hin = H // 2 if ups else H
        xs = [torch.randn(n, hin, hin, ci, device='cuda', generator=g) for _ in range(ROT)]
        u = ops.wino_transform_weights(torch.randn(3, 3, co, ci, device='cuda', generator=g) * 0.1)
        b = torch.randn(co, device='cuda', generator=g)
        if L['fn'] == 'conv2d_wino_pixelnorm':
            fn = lambda i: ops.conv2d_wino_pixelnorm(xs[i % ROT], u, b, n, H, H, 0.37, 0.2, 1e-8, ups=ups)
        elif L['fn'] == 'conv2d_wino_pnbwd':
            pool = kind.endswith('pool')
            ho = H // 2 if pool else H
            ys = torch.randn(n, ho, ho, co, device='cuda', generator=g)
            rs = torch.rand(n * ho * ho, device='cuda', generator=g) + 0.5
            fn = lambda i: ops.conv2d_wino_pnbwd(xs[i % ROT], u, ys, rs, n, H, H, 0.37, 0.2, pool=pool, a=4.0)
        else:
            parts = kind.split('+')
            kw = dict(ups=ups)
            m32 = torch.randn(n, H, H, co, device='cuda', generator=g) if 'mask32' in parts else None
            mb = (torch.randn(n, H, H, co // 4, device='cuda', generator=g) > 0).to(torch.uint8) * 5 if 'maskb' in parts else None
            if m32 is not None or mb is not None:
                kw.update(mask=m32 if m32 is not None else mb, mask_slope=0.2)
            if 'signs' in parts: kw['signs_out'] = True
            for p_ in parts:
                if p_.startswith('pool'):
                    kw['pool'] = True
                    kw['pool_only'] = p_.startswith('poolonly')
            if 'bytes' in parts: kw['y_bytes'] = True
            if 'blend' in parts: kw.update(other=torch.randn(n, H // 2, H // 2, co, device='cuda', generator=g), a=0.6, b=0.4)
            for p_ in parts:
                if p_.startswith('unpool'):
                    um = None
                    if p_ == 'unpoolb': um = (torch.randn(n, 2 * H, 2 * H, co // 4, device='cuda', generator=g) > 0).to(torch.uint8) * 5
                    if p_ == 'unpool32': um = torch.randn(n, 2 * H, 2 * H, co, device='cuda', generator=g)
                    kw.update(unpool=True, upmask=um, up_mul=0.7, mask_slope=0.2)
            bias = b if 'bias' in parts else None
            slope = 0.2 if bias is not None else 1.0
            fn = lambda i: ops.conv2d_wino(xs[i % ROT], u, bias, n, H, H, 0.37, slope, **kw)
        fn(0)                                                  # (first touch: code object, K-slice scratch)
        torch.cuda.synchronize()
        for i in range(REPS):
            fn(i + 1)
        torch.cuda.synchronize()
        L['kernel'] = lib.pg_debug_last_wino_kernel().decode()
        del xs


def make_table(list_json, counters_csv, out_csv):
    layers = json.load(open(list_json))
    per = collections.OrderedDict()
    with open(counters_csv) as f:
        for row in csv.DictReader(f):
            name = row['Kernel_Name']
            if 'conv_wino2_kernel' not in name and 'conv_wino_strip_kernel' not in name and 'conv_wino_kernel' not in name:
                continue
            d = per.setdefault(int(row['Dispatch_Id']), dict(name=name, ns=int(row['End_Timestamp']) - int(row['Start_Timestamp'])))
            d[row['Counter_Name']] = d.get(row['Counter_Name'], 0.0) + float(row['Counter_Value'])
    disp = [per[k] for k in sorted(per)]
    # replay order: per layer 1 first-touch launch + REPS timed ones; the recording step's own launches come first: align from the END
    need = len(layers) * (REPS + 1)
    if len(disp) < need:
        raise SystemExit('only %d conv dispatches in %s, %d expected' % (len(disp), counters_csv, need))
    disp = disp[len(disp) - need:]
    rows = []
    for i, L in enumerate(layers):
        ds = disp[i * (REPS + 1) + 1:(i + 1) * (REPS + 1)]
        us = sum(d['ns'] for d in ds) / len(ds) / 1e3
        busy = 100.0 * sum(d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) for d in ds) / (sum(d['ns'] for d in ds) * 2.4 * 1024)
        nm, nv = sum(d.get('SQ_INSTS_MFMA', 0.0) for d in ds), sum(d.get('SQ_INSTS_VALU', 0.0) for d in ds)
        fl = 2.0 * L['n'] * L['H'] * L['H'] * L['cin'] * L['cout'] * 9
        strip = 'strip' in L['kernel']
        tiles = L['n'] * (L['H'] // 2) ** 2
        ksp = ', true, ' in L['kernel']
        base_wg = -(-tiles // 64) * -(-L['cout'] // 16)
        rows.append([L['kernel'], L['kind'], L['n'], L['H'], L['cin'], L['cout'], L['launches_per_step'], L['cin'] // 8,
                     'one round of resident workgroups' if strip else ('%d x K slices' % base_wg if ksp else str(base_wg)),
                     '%.1f' % us, '%.1f' % busy, '%.2f' % (nv / nm - 1.0 if nm else 0.0), '%.1f' % (fl / us / 1e6), '%.1f' % (fl * 16 / 36 / us / 1e6),
                     '%.3f' % (L['launches_per_step'] * us / 1e3)])
    rows.sort(key=lambda r: -float(r[-1]))
    with open(out_csv, 'w') as f:
        w = csv.writer(f)
        w.writerow(['kernel', 'epilogue form', 'n', 'H=W', 'Cin', 'Cout', 'launches_per_step', 'K_chunks', 'workgroups', 'avg_us_alone_cold',
                    'mfma_busy_pct', 'valu_per_mfma', 'algorithmic_TF', 'executed_TF', 'ms_per_step_if_alone'])
        w.writerows(rows)
    tot = sum(float(r[-1]) for r in rows)
    wb = sum(float(r[-1]) * float(r[10]) for r in rows) / tot
    print('%d distinct Winograd conv launches, %.2f ms per step if each ran alone, time-weighted MFMA-busy %.1f %%' % (len(rows), tot, wb))


if __name__ == '__main__':
    if sys.argv[1] == 'run':
        layers = record_step()
        replay(layers)
        json.dump(layers, open(sys.argv[2], 'w'), indent=0)
        print('%d distinct launches' % len(layers))
    else:
        make_table(sys.argv[2], sys.argv[3], sys.argv[4])
