"""Row-streaming conv (csrc/conv_strip.hip) vs the tile kernel it replaces, on the 1024^2 layer shapes of the schedule.
Interleaved A/B in one process (pg_debug_set_tuning(3, 20) = tile kernel); PG_STRIP_SEG / PG_STRIP_WREG select the
strip variants (read once per process: run the script once per setting)."""
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


def case(n, ci, co, kind):
    g = torch.Generator(device='cuda').manual_seed(1)
    ups = kind == 'ups+pn'
    hin = H // 2 if ups else H
    x = torch.randn(n, hin, hin, ci, device='cuda', generator=g)
    w = torch.randn(3, 3, co, ci, device='cuda', generator=g) * 0.2
    b = torch.randn(co, device='cuda', generator=g)
    m = (torch.randn(n, H, H, co // 4, device='cuda', generator=g) > 0).to(torch.uint8) * 5
    if kind == 'fwd':
        fn = lambda: ops.conv2d(x, w, b, n, H, H, 3, 1, 0.37, slope=0.2)
    elif kind == 'fwd+signs':
        fn = lambda: ops.conv2d(x, w, b, n, H, H, 3, 1, 0.37, slope=0.2, signs_out=True)
    elif kind == 'masked':
        fn = lambda: ops.conv2d(x, w, None, n, H, H, 3, 1, 0.37, mask=m, mask_slope=0.2)
    elif kind in ('pn', 'ups+pn'):
        fn = lambda: ops.conv2d_pixelnorm(x, w, b, n, H, H, 3, 1, 0.37, 0.2, 1e-8, ups=ups)
    res = []
    for rnd in range(3):
        lib.pg_debug_set_tuning(3, -1)
        a = timeit(fn)
        ka = lib.pg_debug_last_conv_kernel().decode()
        lib.pg_debug_set_tuning(3, 20)
        t = timeit(fn)
        lib.pg_debug_set_tuning(3, -1)
        res.append((a, t))
    a, t = min(r[0] for r in res), min(r[1] for r in res)
    fl = 2.0 * n * H * H * ci * co * 9
    byt = n * H * H * 4.0 * (ci / (4 if ups else 1) + co) + (n * H * H * co / 4 if kind in ('masked', 'fwd+signs') else 0)
    print('%-10s n%d %2d->%2d @%d  strip %7.1f us (%5.1f TF, %4.2f TB/s)   tile %7.1f us   x%.2f   [%s]' % (
        kind, n, ci, co, H, a, fl / a / 1e6, byt / a / 1e6, t, t / a, ka), flush=True)


print('PG_STRIP_SEG=%s PG_STRIP_WREG=%s' % (os.environ.get('PG_STRIP_SEG'), os.environ.get('PG_STRIP_WREG')))
for n in (3, 9):
    for ci, co, kind in ((8, 8, 'fwd'), (8, 8, 'masked'), (8, 8, 'fwd+signs'), (8, 8, 'pn'), (16, 8, 'masked'), (16, 8, 'ups+pn')):
        if kind == 'ups+pn' and n == 9:
            continue
        case(n, ci, co, kind)
