"""8 -> 16 conv at 1024^2 (DBlock c2 of the last stage): row-streaming kernel with 16 couts (PG_STRIP_C16=1, direct entry point)
against the Winograd kernel, forward + pool with sign bytes out and the masked + pooled tangent form; inputs rotated (cold cache).
    PG_STRIP_C16=1 python tools/bench_strip16.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
pg = importlib.import_module('pggan-pytorch_amd')
ops, lib = pg.ops, pg._lib.load()
REPS, SETS = 20, 3


def timed(fn):
    for i in range(3):
        fn(i)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for i in range(REPS):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / REPS * 1e3


for N in (9, 3):
    H, ci, co = 1024, 8, 16
    xs = [torch.randn(N, H, H, ci, device='cuda') for _ in range(SETS)]
    w = torch.randn(3, 3, co, ci, device='cuda') * 0.2
    b = torch.randn(co, device='cuda')
    u = ops.wino_transform_weights(w)
    mb = (torch.rand(N, H, H, co // 4, device='cuda') * 16).to(torch.uint8)
    flop = 2.0 * N * H * H * ci * co * 9
    yd, pd = ops.conv2d_pool(xs[0], w, b, N, H, H, 3, 1, 0.5, slope=0.2, y_bytes=True)
    kd = lib.pg_debug_last_conv_kernel().decode()
    yw, pw = ops.conv2d_wino(xs[0], u, b, N, H, H, 0.5, 0.2, pool=True, y_bytes=True)
    print('n%d pooled output rel err %.1e, sign bytes equal %.6f (%s)' % (N, float((pd - pw).norm() / pw.norm()), float((yd == yw).float().mean()), kd))
    t1 = timed(lambda i: ops.conv2d_pool(xs[i % SETS], w, b, N, H, H, 3, 1, 0.5, slope=0.2, y_bytes=True))
    t2 = timed(lambda i: ops.conv2d_wino(xs[i % SETS], u, b, N, H, H, 0.5, 0.2, pool=True, y_bytes=True))
    t3 = timed(lambda i: ops.conv2d_pool(xs[i % SETS], w, None, N, H, H, 3, 1, 0.5, mask=mb, mask_slope=0.2, pool_only=True))
    t4 = timed(lambda i: ops.conv2d_wino(xs[i % SETS], u, None, N, H, H, 0.5, mask=mb, mask_slope=0.2, pool=True, pool_only=True))
    print('n%d @1024 8->16 fwd+pool+bytes: direct %.1f us (%.0f TF)  winograd %.1f us (%.0f TF) | masked+pool: direct %.1f us  winograd %.1f us' % (
        N, t1, flop / t1 * 1e-6, t2, flop / t2 * 1e-6, t3, t4), flush=True)
