"""Sweep conv tile candidate x split-K on the low-parallelism (minibatch 3) shapes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import pggan_amd as pg
ops, lib = pg.ops, pg._lib.load()
SHAPES = [(3, 16, 512, 512), (3, 32, 256, 256), (3, 32, 512, 256), (3, 64, 128, 128), (3, 8, 512, 512), (9, 8, 512, 512), (9, 4, 512, 512), (3, 4, 512, 512), (9, 16, 512, 512)]
def run(f, reps=10):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
for (N, H, ci, co) in SHAPES:
    x = torch.randn(N, H, H, ci, device='cuda'); w = torch.randn(3, 3, co, ci, device='cuda') * 0.05; b = torch.randn(co, device='cuda')
    y = torch.empty(N, H, H, co, device='cuda')
    fl = 2.0 * N * H * H * ci * co * 9
    print('conv n%d @%d %d->%d' % (N, H, ci, co))
    for c in [-1, 1, 2, 3, 4, 5, 6, 7]:
        line = '   cand %2d:' % c
        for ks in ([-1] if c < 0 else [1, 2, 3, 4, 6, 8]):
            lib.pg_debug_set_tuning(0, c); lib.pg_debug_set_tuning(2, ks)
            try:
                ops.conv2d(x, w, b, N, H, H, 3, 1, 0.5, 0.2, out=y)
            except RuntimeError:
                line += '  k%d unsup' % ks
                continue
            t = run(lambda: ops.conv2d(x, w, b, N, H, H, 3, 1, 0.5, 0.2, out=y))
            line += '  k%d %.1fus %.0fTF' % (ks, t * 1e6, fl / t / 1e12)
        print(line + '   ' + lib.pg_debug_last_conv_kernel().decode().replace('conv_igemm_kernel', ''), flush=True)
    lib.pg_debug_set_tuning(0, -1); lib.pg_debug_set_tuning(2, -1)
