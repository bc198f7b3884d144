"""Offline: replay the conv tile cost model against profiles/r01_conv_tile_sweep.txt and report regret."""
import re, sys, math, itertools
rows = []
for l in open('profiles/r01_conv_tile_sweep.txt'):
    m = re.match(r'n(\d+)\s+@(\d+)\s+(\d+)->(\d+)\s+auto\s+([\d.]+)us', l)
    if not m: continue
    N, H, ci, co = [int(m.group(i)) for i in range(1, 5)]
    t = {int(a): float(b) for a, b in re.findall(r'c(\d):(\d+)', l.split('|')[1])}
    rows.append((N, H, ci, co, float(m.group(5)), t))
CANDS = [(256, 16), (128, 64), (128, 32), (128, 16), (64, 64), (64, 32), (64, 16), (16, 64)]
VGPR = [160, 200, 128, 100, 164, 100, 68, 92]
def geom(N, H, bpx):
    TW = min(H, 32, bpx)
    while TW > 4 and bpx // TW < 4 and H >= 4: TW //= 2
    TH = min(bpx // TW, H)
    TN = bpx // (TW * TH)
    ntiles = -(-N // TN) * (H // TH) * (H // TW)
    return TN, TH, TW, ntiles
def model(N, H, ci, co, P):
    ovh_chunk, fixed, eff, atom, splitpen = P
    VEC = 4 if ci % 16 == 0 else 2
    nchunks = ci // (4 * VEC)
    out = {}
    for i, (bpx, bco) in enumerate(CANDS):
        if bpx == 256 and co > 16: continue
        if bco > 16 and co <= 16: continue
        if bco > 32 and co <= 32: continue
        TN, TH, TW, ntiles = geom(N, H, bpx)
        halo = TN * (TH + 2) * (TW + 2)
        kcp = 24 if VEC == 4 else (12 if VEC == 2 else 8)
        lds = (9 * bco + halo) * kcp * 4
        r = max(1, min(160 * 1024 // lds, 512 // VGPR[i], 8))
        blocks = ntiles * (-(-co // bco))
        ks = 1
        if blocks < 192 and nchunks >= 4:
            ks = min(-(-512 // blocks), nchunks)
            cper = -(-nchunks // ks); ks = -(-nchunks // cper)
        cper = -(-nchunks // ks)
        wgs = blocks * ks
        mfma_chunk = (bpx // 16) * (bco // 16) / 4.0 * VEC * 9 * 32.0
        mfma_wg = mfma_chunk * cper
        L = cper * (mfma_chunk + ovh_chunk) + fixed + (atom * bpx * bco + 2000 if ks > 1 else 0)
        full, rem = divmod(wgs, 256 * r)
        T = full * max(L, r * mfma_wg / eff)
        if rem:
            rr = -(-rem // 256)
            T += max(L, rr * mfma_wg / eff)
        out[i] = T + (splitpen if ks > 1 else 0) + 1e-3 * i
    return out
def regret(P, verbose=False):
    tot_auto = tot_model = tot_best = 0
    for (N, H, ci, co, tauto, t) in rows:
        mo = model(N, H, ci, co, P)
        pick = min(mo, key=mo.get)
        best = min(t, key=t.get)
        tot_auto += tauto; tot_model += t[pick]; tot_best += t[best]
        if verbose:
            print('n%-2d @%-4d %3d->%-3d pick c%d %5.0f best c%d %5.0f auto %5.0f  model us %s' % (N, H, ci, co, pick, t[pick], best, t[best], tauto,
                  ' '.join('c%d:%.0f' % (k, v / 2400) for k, v in sorted(mo.items()))))
    return tot_auto, tot_model, tot_best
best = None
for ovh in (300, 500, 700, 1000, 1500):
    for fixed in (1500, 2500, 4000, 6000):
        for eff in (0.7, 0.8, 0.9, 1.0):
            for atom in (2.0, 5.0):
              for sp in (0, 8000, 16000, 24000):
                P = (ovh, fixed, eff, atom, sp)
                r = regret(P)
                if best is None or r[1] < best[0]: best = (r[1], P, r)
print(best)
regret(best[1], True)
