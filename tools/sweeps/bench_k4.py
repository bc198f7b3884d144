"""The two 4x4 layers (latent -> 4x4 of the generator, 4x4 -> 1x1 of the discriminator) in isolation: 16 * Cout * Cin weights
streamed against N samples.    python tools/sweeps/bench_k4.py [reps]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
pg = importlib.import_module('pggan-pytorch_amd')
ops, lib = pg.ops, pg._lib.load()
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 100


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


for N in (3, 6, 9, 16, 64):
    for ci, co in ((512, 512), (256, 256)):
        w = torch.randn(4, 4, co, ci, device='cuda') * 0.1
        bias = torch.randn(co, device='cuda')
        mb = 16 * ci * co * 4 / 1e6
        x1 = torch.randn(N, 1, 1, ci, device='cuda')
        t = timed(lambda: ops.conv2d(x1, w, bias, N, 1, 1, 4, 3, 0.5, 0.2))
        k1 = lib.pg_debug_last_conv_kernel().decode()
        x4 = torch.randn(N, 4, 4, ci, device='cuda')
        t2 = timed(lambda: ops.conv2d(x4, w, bias, N, 4, 4, 4, 0, 0.5, 0.2))
        k2 = lib.pg_debug_last_conv_kernel().decode()
        print('n%-2d %d->%d (%.1f MB of weights): 1x1->4x4 %6.1f us %5.2f TB/s (%s) | 4x4->1x1 %6.1f us %5.2f TB/s (%s)'
              % (N, ci, co, mb, t, mb / t, k1, t2, mb / t2, k2), flush=True)
