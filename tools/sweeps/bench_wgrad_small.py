"""Weight gradient of the 512-channel 3x3 layers on 4x4 / 8x8 maps (K = pixels: 48 .. 576) under the tile-range split
(pg_debug_set_tuning(2, chunks)) and the block shapes of the sweep switch (pg_debug_set_tuning(1, cfg)).
    python tools/sweeps/bench_wgrad_small.py [reps]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
pg = importlib.import_module('pggan-pytorch_amd')
ops, lib = pg.ops, pg._lib.load()
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 50


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


for N, H, ci, co in ((3, 8, 512, 512), (9, 8, 512, 512), (3, 4, 512, 512), (6, 4, 528, 512), (3, 8, 256, 512)):
    x, gz = torch.randn(N, H, H, ci, device='cuda'), torch.randn(N, H, H, co, device='cuda')
    dw, db = torch.zeros(3, 3, co, ci, device='cuda'), torch.zeros(co, device='cuda')
    row = []
    for cfg in (-1, 1, 3):
        for chunks in (-1, 2, 3, 4):
            lib.pg_debug_set_tuning(1, cfg)
            lib.pg_debug_set_tuning(2, chunks)
            t = timed(lambda: ops.conv2d_wgrad(x, gz, dw, db, N, H, H, 3, 1, 0.5))
            row.append('cfg%2d x%2d %5.1f' % (cfg, chunks, t))
    lib.pg_debug_set_tuning(1, -1)
    lib.pg_debug_set_tuning(2, -1)
    print('n%d @%d %d->%d (%s): ' % (N, H, ci, co, lib.pg_debug_last_conv_kernel().decode()[:40]) + ' | '.join(row), flush=True)
