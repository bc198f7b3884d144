import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import pggan_amd as pg
ops, lib = pg.ops, pg._lib.load()
SHAPES = [(3, 8, 512, 512), (9, 8, 512, 512), (3, 16, 512, 512), (3, 4, 512, 512), (3, 32, 256, 256), (3, 64, 128, 128), (3, 128, 64, 64), (16, 4, 512, 512), (16, 8, 512, 512), (48, 4, 512, 512)]
def run(f, reps=10):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
for (N, H, ci, co) in SHAPES:
    x = torch.randn(N, H, H, ci, device='cuda'); gz = torch.randn(N, H, H, co, device='cuda')
    dw = torch.zeros(3, 3, co, ci, device='cuda'); db = torch.zeros(co, device='cuda')
    fl = 2.0 * N * H * H * ci * co * 9
    line = 'wgrad n%d @%d %d->%d:' % (N, H, ci, co)
    for cfg in (0, 3):
        for ch in (-1, 2, 3, 4, 6, 8, 12):
            lib.pg_debug_set_tuning(1, cfg); lib.pg_debug_set_tuning(2, ch)
            try:
                t = run(lambda: ops.conv2d_wgrad(x, gz, dw, db, N, H, H, 3, 1, 0.5))
            except RuntimeError:
                continue
            line += '  c%d/k%d %.1fus' % (cfg, ch, t * 1e6)
    lib.pg_debug_set_tuning(1, -1); lib.pg_debug_set_tuning(2, -1)
    print(line, flush=True)
