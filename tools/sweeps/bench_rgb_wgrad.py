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
for (N, H, C) in [(16, 4, 512), (48, 4, 512), (16, 8, 512), (48, 8, 512), (16, 16, 512), (48, 16, 512), (16, 32, 256), (48, 32, 256), (16, 64, 128), (9, 1024, 8), (3, 1024, 8), (3, 512, 16), (3, 256, 32)]:
    gz = torch.randn(N, H, H, C, device='cuda'); img = torch.randn(N, 3, H, H, device='cuda')
    dw = torch.zeros(C, 3, device='cuda'); db = torch.zeros(C, device='cuda')
    x = torch.randn(N, H, H, C, device='cuda'); g = torch.randn(N, 3, H, H, device='cuda'); dw2 = torch.zeros(3, C, device='cuda'); db2 = torch.zeros(3, device='cuda')
    t1 = run(lambda: ops.fromrgb_wgrad(gz, img, dw, db, N, 3, H, H, 0.5))
    t2 = run(lambda: ops.torgb_wgrad(g, x, dw2, db2, N, 3, H, H, 0.5, 1.0))
    print('n%d @%d C%d: fromrgb_wgrad %.1fus  torgb_wgrad %.1fus' % (N, H, C, t1 * 1e6, t2 * 1e6), flush=True)
