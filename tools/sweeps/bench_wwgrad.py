"""Winograd weight gradient (pg_conv2d_wgrad_wino_nhwc) on the layers of the 1024x1024 schedule, n = 12 (D's batched sweep +
tangent term) and n = 3: microseconds and algorithmic TFLOP/s per launch, inputs rotated over several sets (cold Infinity
Cache).  A/B two builds with PGGAN_HIP_LIB.    python tools/sweeps/bench_wwgrad.py [reps]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
pg = importlib.import_module('pggan-pytorch_amd')
ops, lib = pg.ops, pg._lib.load()
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 30
SETS = int(os.environ.get('SETS', '4'))
CASES = [(12, 16, 512, 512), (12, 32, 256, 512), (12, 64, 128, 256), (12, 128, 64, 128), (12, 256, 32, 64), (12, 128, 64, 64), (12, 64, 128, 128),
         (12, 256, 32, 32), (3, 64, 128, 256), (3, 32, 512, 256), (12, 512, 16, 32), (3, 16, 512, 512), (12, 512, 16, 16), (3, 512, 32, 16), (3, 512, 16, 16), (3, 32, 256, 256), (12, 16, 512, 512)]
for N, H, ci, co in CASES:
    xs = [torch.randn(N, H, H, ci, device='cuda') for _ in range(SETS)]
    gs = [torch.randn(N, H, H, co, device='cuda') for _ in range(SETS)]
    dw, db = torch.zeros(3, 3, co, ci, device='cuda'), torch.zeros(co, device='cuda')
    for i in range(3):
        ops.conv2d_wgrad_wino(xs[i % SETS], gs[i % SETS], dw, db, N, H, H, 0.5)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for i in range(REPS):
        ops.conv2d_wgrad_wino(xs[i % SETS], gs[i % SETS], dw, db, N, H, H, 0.5)
    b.record()
    torch.cuda.synchronize()
    t = a.elapsed_time(b) / REPS * 1e3
    print('wgrad n%-2d @%-3d %3d->%-3d %-34s %7.1f us %6.1f TF' % (N, H, ci, co, lib.pg_debug_last_wino_wgrad_kernel().decode(), t,
                                                                  2.0 * N * H * H * ci * co * 9 / t * 1e-6), flush=True)
