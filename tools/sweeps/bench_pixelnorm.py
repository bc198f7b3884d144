import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import pggan_amd as pg
ops = pg.ops
def run(f, reps=10):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
for (N, H, C) in [(3, 1024, 8), (3, 512, 16), (3, 256, 32), (3, 128, 64), (3, 64, 128)]:
    x = torch.randn(N, H, H, C, device='cuda'); g = torch.randn(N, H, H, C, device='cuda')
    y, r = ops.pixelnorm_fwd(x)
    tf = run(lambda: ops.pixelnorm_fwd(x))
    tb = run(lambda: ops.pixelnorm_lrelu_bwd(g, y, r, 0.2))
    b = 4.0 * N * H * H * C
    print('pixelnorm n%d @%d C%d: fwd %.1fus %.2fTB/s   bwd %.1fus %.2fTB/s' % (N, H, C, tf * 1e6, 2 * b / tf / 1e12, tb * 1e6, 3 * b / tb / 1e12), flush=True)
