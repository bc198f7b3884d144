"""Row-streaming Winograd conv (csrc/conv_wino_strip.hip) vs the tile kernel (conv_wino2_kernel) on the thin 3x3 layers of the
1024^2 / 512^2 / 256^2 stages, with the epilogues the train step uses.  Interleaved A/B in one process
(pg_debug_set_wino(20) = tile kernels only); inputs rotate over several buffers so that every launch reads cold data
(a single hot input would sit in the 256 MB Infinity Cache)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pggan_amd as pg  # noqa: E402

ops, lib = pg.ops, pg._lib.load()
ROT = int(os.environ.get('BW_ROT', '4'))
ONLY = os.environ.get('BW_ONLY')


def timeit(fn, reps=16):
    for i in range(ROT):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def case(n, H, ci, co, kind):
    g = torch.Generator(device='cuda').manual_seed(1)
    ups = kind.startswith('ups')
    hin = H // 2 if ups else H
    xs = [torch.randn(n, hin, hin, ci, device='cuda', generator=g) for _ in range(ROT)]
    w = torch.randn(3, 3, co, ci, device='cuda', generator=g) * 0.2
    u = ops.wino_transform_weights(w)
    b = torch.randn(co, device='cuda', generator=g)
    mb = (torch.randn(n, H, H, co // 4, device='cuda', generator=g) > 0).to(torch.uint8) * 5
    outb = n * H * H * co * 4.0
    if kind == 'fwd':
        fn = lambda i: ops.conv2d_wino(xs[i % ROT], u, b, n, H, H, 0.37, 0.2)
    elif kind == 'fwd+signs':
        fn = lambda i: ops.conv2d_wino(xs[i % ROT], u, b, n, H, H, 0.37, 0.2, signs_out=True)
        outb *= 1 + 1 / 16
    elif kind == 'maskb':
        fn = lambda i: ops.conv2d_wino(xs[i % ROT], u, None, n, H, H, 0.37, mask=mb, mask_slope=0.2)
        outb *= 1 + 1 / 16
    elif kind == 'pool+bytes':
        fn = lambda i: ops.conv2d_wino(xs[i % ROT], u, b, n, H, H, 0.37, 0.2, pool=True, y_bytes=True)
        outb = outb / 4 + outb / 16
    elif kind == 'maskb+pool':
        fn = lambda i: ops.conv2d_wino(xs[i % ROT], u, None, n, H, H, 0.37, mask=mb, mask_slope=0.2, pool=True, pool_only=True)
        outb = outb / 4 + outb / 16
    elif kind == 'poolonly':
        fn = lambda i: ops.conv2d_wino(xs[i % ROT], u, None, n, H, H, 0.37, pool=True, a=4.0, pool_only=True)
        outb = outb / 4
    elif kind == 'unpool':
        um = (torch.randn(n, 2 * H, 2 * H, co // 4, device='cuda', generator=g) > 0).to(torch.uint8) * 5
        fn = lambda i: ops.conv2d_wino(xs[i % ROT], u, None, n, H, H, 0.37, mask_slope=0.2, unpool=True, upmask=um, up_mul=0.7)
        outb = outb * 4 * (1 + 1 / 16)
    elif kind in ('pn', 'ups+pn'):
        fn = lambda i: ops.conv2d_wino_pixelnorm(xs[i % ROT], u, b, n, H, H, 0.37, 0.2, 1e-8, ups=ups)
    elif kind in ('pnbwd', 'pnbwd+pool'):
        pool = kind.endswith('pool')
        ho = H // 2 if pool else H
        ys = torch.randn(n, ho, ho, co, device='cuda', generator=g)
        rs = torch.rand(n * ho * ho, device='cuda', generator=g) + 0.5
        fn = lambda i: ops.conv2d_wino_pnbwd(xs[i % ROT], u, ys, rs, n, H, H, 0.37, 0.2, pool=pool, a=4.0)
        outb = (outb / 4 if pool else outb) * 2
    else:
        raise ValueError(kind)
    res = []
    for _ in range(3):
        lib.pg_debug_set_wino(0)
        a = timeit(fn)
        ka = lib.pg_debug_last_wino_kernel().decode()
        lib.pg_debug_set_wino(20)
        t = timeit(fn)
        kt = lib.pg_debug_last_wino_kernel().decode()
        lib.pg_debug_set_wino(0)
        res.append((a, t))
    a, t = min(r[0] for r in res), min(r[1] for r in res)
    fl = 2.0 * n * H * H * ci * co * 9
    byt = n * hin * hin * ci * 4.0 + outb
    print('%-11s n%d %2d->%2d @%-4d strip %7.1f us (%5.1f TF, %4.2f TB/s)  tile %7.1f us  x%.2f  [%s | %s]' % (
        kind, n, ci, co, H, a, fl / a / 1e6, byt / a / 1e6, t, t / a, ka, kt), flush=True)


CASES = [
    # 1024^2 stage: D c2 (8->16) forward with pool + sign bytes, tangent (mask bytes), G backward-data of c1 (pn adjoint + pool)
    (1024, 8, 16, 'pool+bytes'), (1024, 8, 16, 'maskb+pool'), (1024, 8, 16, 'pnbwd+pool'), (1024, 8, 16, 'fwd'),
    # 512^2 stage
    (512, 16, 16, 'fwd+signs'), (512, 16, 16, 'maskb'), (512, 16, 32, 'pool+bytes'), (512, 32, 16, 'maskb'), (512, 32, 16, 'unpool'),
    (512, 16, 16, 'pn'), (512, 32, 16, 'ups+pn'), (512, 16, 16, 'pnbwd'), (512, 16, 32, 'pnbwd+pool'), (512, 16, 32, 'maskb+pool'), (512, 16, 16, 'unpool'),
    # 256^2 stage
    (256, 32, 32, 'fwd+signs'), (256, 32, 32, 'maskb'), (256, 32, 64, 'pool+bytes'), (256, 32, 32, 'pn'), (256, 32, 64, 'maskb+pool'), (256, 32, 32, 'unpool'),
]
print('PG_WSTRIP_WINO_SEG=%s PG_WSTRIP_WINO_NCB=%s' % (os.environ.get('PG_WSTRIP_WINO_SEG'), os.environ.get('PG_WSTRIP_WINO_NCB')))
for n in (9, 3):
    for H, ci, co, kind in CASES:
        if ONLY and ONLY not in ('%d:%d:%d:%s' % (H, ci, co, kind)):
            continue
        if n == 9 and ('pn' in kind):
            continue                                           # generator-side epilogues: 3 images only
        case(n, H, ci, co, kind)
