"""Persistent Winograd conv (conv_wino2p_kernel, pg_debug_set_wino_pers) against the one-block-per-workgroup kernel on the
few-chunk layers of the 256^2 .. 1024^2 stages.    python tools/bench_pers.py [reps]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
pg = importlib.import_module('pggan-pytorch_amd')
ops, lib = pg.ops, pg._lib.load()
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 30
CASES = [(3, 512, 16, 16), (9, 512, 16, 16), (3, 256, 32, 32), (9, 256, 32, 32), (9, 1024, 8, 16), (3, 1024, 8, 16), (9, 512, 16, 32), (9, 512, 32, 16),
         (3, 512, 32, 16), (9, 256, 32, 64), (3, 128, 64, 64), (9, 128, 64, 64)]


def timed(fn):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(REPS):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / REPS * 1e3


SETS = int(os.environ.get('SETS', '1'))          # > 1: rotate over that many input / output sets (cold Infinity Cache: 256 MB)
for N, H, ci, co in CASES:
    xs = [torch.randn(N, H, H, ci, device='cuda') for _ in range(SETS)]
    outs = [torch.empty(N, H, H, co, device='cuda') for _ in range(SETS)]
    x = xs[0]
    turn = [0]

    def rot():
        turn[0] = (turn[0] + 1) % SETS
        return xs[turn[0]], outs[turn[0]]
    w = torch.randn(3, 3, co, ci, device='cuda') * 0.1
    bias = torch.randn(co, device='cuda')
    mask = (torch.rand(N, H, H, co // 4, device='cuda') * 16).to(torch.uint8)
    u = ops.wino_transform_weights(w)
    flop = 2.0 * N * H * H * ci * co * 9
    row, ref = [], None
    for mode in (0, 2):
        lib.pg_debug_set_wino_pers(mode)
        y = ops.conv2d_wino(x, u, bias, N, H, H, 0.5, 0.2)
        ym = ops.conv2d_wino(x, u, None, N, H, H, 0.5, mask=mask, mask_slope=0.2)
        if ref is None:
            ref = (y, ym)
        else:
            assert torch.equal(y, ref[0]) and torch.equal(ym, ref[1]), 'persistent kernel differs'
        t = timed(lambda: (lambda xo: ops.conv2d_wino(xo[0], u, bias, N, H, H, 0.5, 0.2, out=xo[1]))(rot()))
        name = lib.pg_debug_last_wino_kernel().decode()
        tm = timed(lambda: (lambda xo: ops.conv2d_wino(xo[0], u, None, N, H, H, 0.5, mask=mask, mask_slope=0.2, out=xo[1]))(rot()))
        row.append('%-28s fwd %6.1f us %5.1f TF | masked %6.1f us %5.1f TF' % (name, t, flop / t * 1e-6, tm, flop / tm * 1e-6))
    lib.pg_debug_set_wino_pers(-1)
    print('n%d @%d %d->%d: ' % (N, H, ci, co) + ' || '.join(row), flush=True)
