"""Winograd conv with the K loop sliced across workgroups (pg_set_workspace) vs the unsplit launch vs the direct kernels, on the
small-map layers of the 1024x1024 schedule (minibatch 3 per GPU: N = 3 in the G step, 9 in D's batched sweep).
    python tools/sweeps/bench_ksplit.py [reps]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
pg = importlib.import_module('pggan-pytorch_amd')
ops, lib = pg.ops, pg._lib.load()
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 50
CASES = [(3, 8, 512, 512), (3, 8, 256, 512), (9, 8, 512, 512), (3, 16, 512, 512), (9, 16, 512, 512), (3, 32, 512, 256), (3, 32, 256, 256), (3, 32, 256, 512),
         (3, 32, 512, 512), (9, 32, 512, 512), (9, 32, 256, 256), (3, 64, 128, 128), (3, 64, 256, 128), (3, 64, 128, 256), (3, 128, 64, 64)]


def timed(fn):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(REPS):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / REPS * 1e3


for N, H, ci, co in CASES:
    x = torch.randn(N, H, H, ci, device='cuda')
    w = torch.randn(3, 3, co, ci, device='cuda') * 0.1
    bias = torch.randn(co, device='cuda')
    u = ops.wino_transform_weights(w)
    flop = 2.0 * N * H * H * ci * co * 9
    row = []
    t = timed(lambda: ops.conv2d(x, w, bias, N, H, H, 3, 1, 0.5, 0.2))
    row.append('direct %6.1f us %5.1f TF (%s)' % (t, flop / t * 1e-6, lib.pg_debug_last_conv_kernel().decode()[5:28]))
    for ks in [int(v) for v in os.environ.get('KS', '0,-1,2,3,4,6,8').split(',')]:
        lib.pg_debug_set_wino_ksplit(ks)
        t = timed(lambda: ops.conv2d_wino(x, u, bias, N, H, H, 0.5, 0.2))
        split = ', true, ' in lib.pg_debug_last_wino_kernel().decode()
        row.append('ks%2d%s %6.1f us %5.1f TF' % (ks, '*' if split else ' ', t, flop / t * 1e-6))
    lib.pg_debug_set_wino_ksplit(-1)
    print('n%d @%d %d->%d: ' % (N, H, ci, co) + ' | '.join(row), flush=True)
