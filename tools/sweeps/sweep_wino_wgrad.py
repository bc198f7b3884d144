"""Winograd weight gradient vs the direct weight-gradient kernels."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import pggan_amd as pg
ops, lib = pg.ops, pg._lib.load()
SHAPES = [(9, 16, 512, 512), (3, 16, 512, 512), (9, 32, 256, 512), (3, 32, 256, 256), (9, 64, 128, 256), (3, 64, 128, 128), (9, 128, 64, 128), (3, 128, 64, 64),
          (9, 256, 32, 64), (3, 256, 32, 32), (9, 512, 16, 32), (3, 512, 16, 32), (9, 512, 32, 16), (3, 512, 32, 16), (9, 512, 16, 16), (3, 512, 16, 16),
          (9, 1024, 8, 16), (3, 1024, 8, 16), (9, 1024, 8, 8), (3, 1024, 8, 8), (3, 1024, 16, 8), (6, 512, 8, 8), (9, 8, 512, 512), (16, 16, 512, 512), (48, 16, 512, 512)]
def run(f, reps=10):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
for (N, H, ci, co) in SHAPES:
    x = torch.randn(N, H, H, ci, device='cuda'); gz = torch.randn(N, H, H, co, device='cuda')
    dw0 = torch.zeros(3, 3, co, ci, device='cuda'); db0 = torch.zeros(co, device='cuda')
    dw1 = torch.zeros(3, 3, co, ci, device='cuda'); db1 = torch.zeros(co, device='cuda')
    fl = 2.0 * N * H * H * ci * co * 9
    ops.conv2d_wgrad(x, gz, dw0, db0, N, H, H, 3, 1, 0.5)
    try:
        ops.conv2d_wgrad_wino(x, gz, dw1, db1, N, H, H, 0.5)
    except RuntimeError as e:
        print('wgrad n%d @%d %d->%d: unsupported' % (N, H, ci, co)); continue
    err = float((dw1 - dw0).abs().max() / dw0.abs().max()); errb = float((db1 - db0).abs().max() / db0.abs().max())
    t0 = run(lambda: ops.conv2d_wgrad(x, gz, dw0, db0, N, H, H, 3, 1, 0.5))
    t1 = run(lambda: ops.conv2d_wgrad_wino(x, gz, dw1, db1, N, H, H, 0.5))
    print('wgrad n%d @%d %d->%d: direct %.1fus %.0fTF   wino %.1fus %.0fTF (%.2fx)   rel err dw %.1e db %.1e' % (
        N, H, ci, co, t0 * 1e6, fl / t0 / 1e12, t1 * 1e6, fl / t1 / 1e12, t0 / t1, err, errb), flush=True)
