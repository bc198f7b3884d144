"""The RGB-side fusions of round 6 at the 1024^2 stage, each against the two launches it replaces, alone on the device:
  forward  c1(fromRGB(img)): pg_fromrgb_fwd + pg_conv2d_nhwc            vs pg_conv2d_fromrgb_nhwc (fromRGB in the conv's gather)
  backward fromRGB^T(c1^T(gz)): pg_conv2d_nhwc (masked) + pg_fromrgb_bwd_data vs pg_conv2d_masked_fromrgb_bwd_nhwc (with / without the 8-channel gradient)
  generator toRGB(pixelnorm(c2(x))): pg_conv2d_pixelnorm_nhwc + pg_torgb_fwd    vs pg_conv2d_pixelnorm_torgb_nhwc
    python tools/sweeps/bench_fromrgb_gather.py [reps]"""
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


for N, H, C in [(3, 1024, 3), (6, 1024, 3), (9, 1024, 3), (3, 512, 3), (8, 256, 1)]:
    img = torch.randn(N, C, H, H, device='cuda')
    rw, rb = torch.randn(8, C, device='cuda'), torch.randn(8, device='cuda')
    w, b = torch.randn(3, 3, 8, 8, device='cuda') * 0.2, torch.randn(8, device='cuda')
    t_rgb = timed(lambda: ops.fromrgb_fwd(img, rw, rb, N, C, H, H, 0.6, 0.2, signs_out=True))
    x0, _ = ops.fromrgb_fwd(img, rw, rb, N, C, H, H, 0.6, 0.2, signs_out=True)
    t_conv = timed(lambda: ops.conv2d(x0, w, b, N, H, H, 3, 1, 0.4, 0.2, signs_out=True))
    k1 = lib.pg_debug_last_conv_kernel().decode()
    t_two = timed(lambda: ops.conv2d(ops.fromrgb_fwd(img, rw, rb, N, C, H, H, 0.6, 0.2, signs_out=True)[0], w, b, N, H, H, 3, 1, 0.4, 0.2, signs_out=True))
    t_one = timed(lambda: ops.conv2d_fromrgb(img, rw, rb, 0.6, 0.2, w, b, N, C, H, H, 0.4, 0.2))
    k2 = lib.pg_debug_last_conv_kernel().decode()
    px = N * H * H
    print('n%d @%d C%d: fromRGB %6.1f us (%.2f TB/s) + conv %6.1f us (%.2f TB/s, %s) = %6.1f back to back | fused %6.1f us (%.2f TB/s, %s)'
          % (N, H, C, t_rgb, px * (4 * C + 34) / t_rgb * 1e-6, t_conv, px * 66 / t_conv * 1e-6, k1[:34], t_two, t_one, px * (4 * C + 36) / t_one * 1e-6, k2), flush=True)

for N, H, C in [(3, 1024, 3), (6, 1024, 3)]:
    gz = torch.randn(N, H, H, 8, device='cuda'); wt = torch.randn(3, 3, 8, 8, device='cuda') * 0.2
    mb = (torch.rand(N, H, H, 2, device='cuda') * 16).to(torch.uint8)
    rw = torch.randn(8, C, device='cuda'); gi = torch.empty(N, C, H, H, device='cuda')
    t1 = timed(lambda: ops.conv2d(gz, wt, None, N, H, H, 3, 1, 0.4, 1.0, mask=mb, mask_slope=0.2))
    gf = ops.conv2d(gz, wt, None, N, H, H, 3, 1, 0.4, 1.0, mask=mb, mask_slope=0.2)
    t2 = timed(lambda: ops.fromrgb_bwd_data(gf, rw, gi, N, C, H, H, 0.6))
    t12 = timed(lambda: ops.fromrgb_bwd_data(ops.conv2d(gz, wt, None, N, H, H, 3, 1, 0.4, 1.0, mask=mb, mask_slope=0.2), rw, gi, N, C, H, H, 0.6))
    t3 = timed(lambda: ops.conv2d_masked_fromrgb_bwd(gz, wt, mb, 0.2, rw, 0.6, N, C, H, H, 0.4, keep_gf=True, gimg=gi))
    t4 = timed(lambda: ops.conv2d_masked_fromrgb_bwd(gz, wt, mb, 0.2, rw, 0.6, N, C, H, H, 0.4, keep_gf=False, gimg=gi))
    print('n%d @%d backward: masked conv %6.1f us + fromRGB adjoint %6.1f us = %6.1f back to back | fused %6.1f us, without the 8-channel gradient %6.1f us' % (N, H, t1, t2, t12, t3, t4), flush=True)
    x = torch.randn(N, H, H, 8, device='cuda'); w = torch.randn(3, 3, 8, 8, device='cuda') * 0.2; b = torch.randn(8, device='cuda')
    tw, tb = torch.randn(C, 8, device='cuda'), torch.randn(C, device='cuda')
    out = torch.empty(N, C, H, H, device='cuda')
    t1 = timed(lambda: ops.conv2d_pixelnorm(x, w, b, N, H, H, 3, 1, 0.4, 0.2, 1e-8))
    y, r = ops.conv2d_pixelnorm(x, w, b, N, H, H, 3, 1, 0.4, 0.2, 1e-8)
    t2 = timed(lambda: ops.torgb_fwd(y, tw, tb, N, C, H, H, 0.7, out=out))
    t12 = timed(lambda: ops.torgb_fwd(ops.conv2d_pixelnorm(x, w, b, N, H, H, 3, 1, 0.4, 0.2, 1e-8)[0], tw, tb, N, C, H, H, 0.7, out=out))
    t3 = timed(lambda: ops.conv2d_pixelnorm_torgb(x, w, b, tw, tb, N, C, H, H, 0.4, 0.2, 0.7, 1e-8, out=out))
    print('n%d @%d generator: conv + PixelNorm %6.1f us + toRGB %6.1f us = %6.1f back to back | fused %6.1f us' % (N, H, t1, t2, t12, t3), flush=True)
