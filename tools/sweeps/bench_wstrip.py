"""Row-streaming weight gradient (csrc/conv_strip.hip) vs the tile kernel, interleaved A/B (pg_debug_set_tuning(1, 20) = tile)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pggan_amd as pg  # noqa: E402

ops, lib = pg.ops, pg._lib.load()
H = int(os.environ.get('BS_H', '1024'))


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print('PG_WSTRIP_SEG=%s' % os.environ.get('PG_WSTRIP_SEG'))
for n in (3, 9):
    for ci, co, ups in ((8, 8, 0), (8, 16, 0), (16, 8, 1)):
        if n == 9 and (ci, co) != (8, 8):
            continue
        g = torch.Generator(device='cuda').manual_seed(1)
        hin = H // 2 if ups else H
        x = torch.randn(n, hin, hin, ci, device='cuda', generator=g)
        gz = torch.randn(n, H, H, co, device='cuda', generator=g)
        dw = torch.zeros(3, 3, co, ci, device='cuda')
        db = torch.zeros(co, device='cuda')
        fn = lambda: ops.conv2d_wgrad(x, gz, dw, db, n, H, H, 3, 1, 0.41, ups=bool(ups))
        res = []
        for rnd in range(3):
            lib.pg_debug_set_tuning(1, -1)
            a = timeit(fn)
            ka = lib.pg_debug_last_conv_kernel().decode()
            lib.pg_debug_set_tuning(1, 20)
            t = timeit(fn)
            lib.pg_debug_set_tuning(1, -1)
            res.append((a, t))
        a, t = min(r[0] for r in res), min(r[1] for r in res)
        fl = 2.0 * n * H * H * ci * co * 9
        byt = n * H * H * 4.0 * (ci / (4 if ups else 1) + co)
        print('wgrad n%d %2d->%2d @%d%s  strip %7.1f us (%5.1f TF, %4.2f TB/s)   tile %7.1f us   x%.2f   [%s]' % (
            n, ci, co, H, ' ups' if ups else '', a, fl / a / 1e6, byt / a / 1e6, t, t / a, ka), flush=True)
